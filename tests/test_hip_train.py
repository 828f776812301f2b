"""GPU tests of the LoRA fine-tune backward kernels against torch autograd (fp32 on the CPU) of the
oracle's forward ops.  Gradients are bf16 tensors: tolerance 1.5% of the reference's max magnitude
(+ 1 bf16 ulp), the level at which two bf16 backward passes agree."""
import math

import pytest
import torch

from conftest import record_parity
from dualhyp_amd.synth import uniform, stream_id

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def U(shape, bound, name, seed=21):
    return uniform(shape, bound, stream_id(seed, name))


def close(got, want, what, rel=1.5e-2):
    got, want = got.float().cpu(), want.float()
    tol = rel * want.abs().max().item() + 1e-6
    err = (got - want).abs().max().item()
    assert err <= tol, f"{what}: max err {err:.4e} > {tol:.4e}"


def test_swiglu_rmsnorm_tn_bwd():
    from dualhyp_amd import ops
    F = torch.nn.functional
    rows, I, d = 37, 384, 256
    g, u, dact = U((rows, I), 2.0, "g"), U((rows, I), 1.0, "u"), U((rows, I), 1.0, "da")
    gf, uf = g.float().requires_grad_(), u.float().requires_grad_()
    (F.silu(gf) * uf).backward(dact.float())
    out = ops.swiglu_bwd(dact.to(DEV), g.to(DEV), u.to(DEV))
    close(out[:, :I], gf.grad, "swiglu dg")
    close(out[:, I:], uf.grad, "swiglu du")
    # rmsnorm
    x, w, dy, dres = U((rows, d), 2.0, "x"), (1 + U((d,), 0.25, "w").float()).bfloat16(), U((rows, d), 1.0, "dy"), U((rows, d), 1.0, "dr")
    xf = x.float().requires_grad_()
    y = w.float() * (xf * torch.rsqrt(torch.mean(xf * xf, -1, keepdim=True) + 1e-5))
    y.backward(dy.float())
    close(ops.rmsnorm_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), 1e-5), xf.grad, "rmsnorm dx")
    close(ops.rmsnorm_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), 1e-5, dres=dres.to(DEV)), xf.grad + dres.float(), "rmsnorm dx + dres")
    # token contraction (LoRA grads), accumulate into fp32
    a, b = U((70, 48), 1.0, "ta"), U((70, 256), 1.0, "tb")
    out = torch.ones((16, 256), dtype=torch.float32, device=DEV)
    ops.tn_accum(a.to(DEV)[:, 16:32], b.to(DEV), out, scale=0.5, accumulate=True)
    want = 1 + 0.5 * a[:, 16:32].float().T @ b.float()
    assert (out.cpu() - want).abs().max().item() <= 1e-4 * want.abs().max().item()
    ops.tn_accum(a.to(DEV)[:, 16:32], b.to(DEV), out, scale=2.0, accumulate=False)
    # packed micro-batches: a long token loop is split over the grid in 512-token chunks, summed in order (both orientations)
    for T in (1025, 4480):
        a2, b2 = U((T, 48), 1.0, f"ta{T}").to(DEV), U((T, 320), 1.0, f"tb{T}").to(DEV)
        for A_, B_ in ((a2[:, 16:32], b2), (b2, a2[:, :48])):
            o = torch.full((A_.size(1), B_.size(1)), 3.0, dtype=torch.float32, device=DEV)
            ops.tn_accum(A_, B_, o, scale=0.25, accumulate=True)
            w = 3.0 + 0.25 * (A_.float().T @ B_.float())
            assert (o - w).abs().max().item() <= 2e-5 * w.abs().max().item() + 1e-4, (T, tuple(o.shape))
            o2 = torch.empty_like(o)
            ops.tn_accum(A_, B_, o2, scale=0.25, accumulate=False)
            assert torch.equal(o2 + 3.0, o) or (o2 + 3.0 - o).abs().max().item() <= 1e-5 * w.abs().max().item()
    assert (out.cpu() - 2 * (want - 1) / 0.5).abs().max().item() <= 1e-3


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_cross_entropy_kernels(golden, dtype):
    """loss.hip: per-row CE with ignore_index -1 and its gradient vs torch fp32 on the CPU, and the
    reference's own chunked_cross_entropy values (tests/golden/misc_ce: both normalisations, Q5)."""
    from dualhyp_amd import ops
    from dualhyp_amd.utils import chunked_cross_entropy
    F = torch.nn.functional
    rows, V = 45, 32000
    lg = (U((rows, V), 6.0, "celg").float() * 1.0).to(dtype)
    tg = torch.randint(0, V, (rows,), generator=torch.Generator().manual_seed(3))
    tg[::4] = -1
    ref_in = lg.float().requires_grad_()
    per_ref = F.cross_entropy(ref_in, tg, ignore_index=-1, reduction="none")
    loss, lse = ops.cross_entropy_fwd(lg.to(DEV), tg.to(DEV))
    assert torch.allclose(loss.cpu(), per_ref.detach(), rtol=2e-6, atol=2e-5)
    assert torch.allclose(lse.cpu(), torch.logsumexp(lg.float(), -1), rtol=2e-6, atol=2e-5)
    assert (loss.cpu()[::4] == 0).all()
    g = torch.rand(rows, generator=torch.Generator().manual_seed(4))
    per_ref.backward(g)
    d = ops.cross_entropy_bwd(lg.to(DEV), tg.to(DEV), lse, g.to(DEV))
    assert d.dtype == dtype and (d.cpu()[::4] == 0).all()
    tol = 1e-6 if dtype == torch.float32 else 2.0 ** -8
    assert ((d.float().cpu() - ref_in.grad).abs() <= tol * ref_in.grad.abs() + 1e-7).all()
    # the reference's normalisations through the public function (autograd node over the kernels)
    t, _ = golden("misc_ce")
    lgt, tgt = t["ce_logits"].to(DEV), t["ce_targets"].to(DEV)
    chunks = list(lgt.split(8, dim=1))
    for key, args in (("ce_list_chunked", (chunks, tgt, 8)), ("ce_list_unchunked", (chunks, tgt, 0)),
                      ("ce_tensor_chunked", (lgt, tgt, 16)), ("ce_tensor_unchunked", (lgt, tgt, 0))):
        got = chunked_cross_entropy(args[0], args[1], chunk_size=args[2])
        assert abs(got.item() - t[key].item()) <= 2e-6 * abs(t[key].item()) + 1e-6, key
    x = lgt.clone().requires_grad_()
    chunked_cross_entropy(x, tgt, chunk_size=0).backward()
    xr = t["ce_logits"].clone().requires_grad_()
    F.cross_entropy(xr.reshape(-1, xr.size(-1)), t["ce_targets"].reshape(-1), ignore_index=-1).backward()
    assert torch.allclose(x.grad.cpu(), xr.grad, rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("hs,n_head,n_groups", [(64, 32, 4), (64, 4, 2), (128, 8, 2)])
def test_rope_and_attention_bwd(hs, n_head, n_groups):
    from dualhyp_amd import ops
    from oracle import ger_oracle as O
    F = torch.nn.functional
    qpk = n_head // n_groups
    lens = [70, 33, 1, 128]
    n_tok, s_max = sum(lens), 128
    width = (n_head + 2 * n_groups) * hs
    qkv = U((n_tok, width), 1.0, f"bq{hs}")
    dout = U((n_tok, n_head * hs), 1.0, f"bdo{hs}")
    cos, sin = O.build_rope_cache(s_max, hs)
    i32 = torch.int32
    slot = torch.cat([torch.full((n,), i, dtype=i32) for i, n in enumerate(lens)]).to(DEV)
    pos = torch.cat([torch.arange(n, dtype=i32) for n in lens]).to(DEV)
    starts = torch.tensor([sum(lens[:i]) for i in range(len(lens))], dtype=i32, device=DEV)
    qlen = torch.tensor(lens, dtype=i32, device=DEV)
    B = len(lens)
    kc = torch.zeros((B, n_groups, s_max, hs), dtype=torch.bfloat16, device=DEV)
    vt = torch.zeros((B, n_groups, hs, s_max), dtype=torch.bfloat16, device=DEV)
    k_out = torch.empty((n_tok, n_groups, hs), dtype=torch.bfloat16, device=DEV)
    v_out = torch.empty_like(k_out)
    q = ops.qkv_rope_cache(qkv.to(DEV), cos.to(DEV), sin.to(DEV), slot, pos, kc, vt, n_head, n_groups, k_out=k_out, v_out=v_out)
    lse = torch.empty((n_tok, n_head), dtype=torch.float32, device=DEV)
    y = ops.attn_prefill(q, kc, vt, torch.arange(B, dtype=i32, device=DEV), starts, qlen, torch.zeros(B, dtype=i32, device=DEV),
                         max(lens), lse=lse)
    dq, dk, dv = ops.attn_bwd(q, k_out, v_out, y, dout.to(DEV), lse, starts, qlen, max(lens))
    # round 4's kernels (dq: one block per group with K / V / K^T staged in LDS; dkdv at hs 64: q / dO tiles through an LDS image with
    # transposed reads) run the same products in the same order as the per-wave kernels of rounds 2-3: the same bits
    from dualhyp_amd import _lib
    try:
        _lib.load().dh_set_tuning(27, 0)
        _lib.load().dh_set_tuning(29, 0)
        dq0, dk0, dv0 = ops.attn_bwd(q, k_out, v_out, y, dout.to(DEV), lse, starts, qlen, max(lens))
    finally:
        _lib.load().dh_set_tuning(27, 1)
        _lib.load().dh_set_tuning(29, 1)
    assert torch.equal(dq, dq0) and torch.equal(dk, dk0) and torch.equal(dv, dv0)
    dqkv = ops.qkv_rope_bwd(dq, dk, dv, cos.to(DEV), sin.to(DEV), pos)
    # reference: fp32 autograd through split + rope + SDPA, per sequence
    t0 = 0
    for n in lens:
        x = qkv[t0:t0 + n].float().requires_grad_()
        x5 = x.view(1, n, n_groups, qpk + 2, hs).permute(0, 2, 3, 1, 4)
        qq, kk, vv = x5.split((qpk, 1, 1), dim=2)
        qq = qq.reshape(1, -1, n, hs)
        kk = kk.expand(1, n_groups, qpk, n, hs).reshape(1, -1, n, hs)
        vv = vv.expand(1, n_groups, qpk, n, hs).reshape(1, -1, n, hs)
        c, s = cos[:n].float(), sin[:n].float()
        qq, kk = O.apply_rope(qq, c, s), O.apply_rope(kk, c, s)
        o = F.scaled_dot_product_attention(qq, kk, vv, is_causal=True, scale=1 / math.sqrt(hs)).transpose(1, 2).reshape(n, -1)
        lse_ref = torch.logsumexp((qq @ kk.transpose(-1, -2) / math.sqrt(hs)).masked_fill(~torch.tril(torch.ones(n, n, dtype=torch.bool)), -1e30), -1)
        assert (lse[t0:t0 + n].cpu() - lse_ref[0].T).abs().max().item() < 2e-2, "forward log-sum-exp"
        o.backward(dout[t0:t0 + n].float())
        close(dqkv[t0:t0 + n], x.grad, f"d(qkv) seq len {n}", rel=2e-2)
        t0 += n


@pytest.mark.parametrize("name", ["tiny_r4", "tiny_hs128_r16"])
def test_train_micro_step_matches_reference(golden, name):
    """One fine-tune micro-step (finetune/ger.py:278-285) through GPT.forward under autograd: loss and
    every LoRA gradient against what the REFERENCE's autograd produced (tests/golden, bf16 run; the fp32
    run of the same weights is the yardstick for what bf16 noise does to these gradients)."""
    from dualhyp_amd import GPT, Config, chunked_cross_entropy
    from dualhyp_amd.synth import synth_state_dict
    from dualhyp_amd.train import prepare_for_training
    t, meta = golden(name)
    cfg = Config(**meta["config"])
    sd = synth_state_dict(cfg, seed=meta["seed"], norm_jitter=meta["norm_jitter"], weight_scale=meta["weight_scale"], device=DEV)
    m = GPT(cfg).to(device=DEV, dtype=torch.bfloat16)
    m.load_state_dict(sd)
    m.cpu_rsqrt_vec_width = 32        # the golden is a CPU run of the reference (Q11)
    m.train()
    params = prepare_for_training(m)
    assert all(p.dtype == torch.float32 and p.requires_grad for p in params)
    idx = torch.stack([t["idx0"], t["idx1"]]).to(DEV)
    labels = t["train_labels"].to(DEV)
    logits = m(idx, lm_head_chunk_size=8)
    assert isinstance(logits, list) and logits[0].requires_grad
    logits[-1] = logits[-1][..., :-1, :]
    loss = chunked_cross_entropy(logits, labels[..., 1:], chunk_size=8)
    (loss / 32).backward()
    ref_loss, f32_loss = t["bf16.train_loss"].float().item(), t["fp32.train_loss"].item()
    assert abs(loss.item() - f32_loss) <= max(1.3 * abs(ref_loss - f32_loss), 1e-2), (loss.item(), ref_loss, f32_loss)   # tiny model: 5.8e-3 measured
    worst = 0.0
    for n, p in m.named_parameters():
        if "lora_" not in n:
            assert p.grad is None
            continue
        g32, gbf = t[f"fp32.grad.{n}"].float(), t[f"bf16.grad.{n}"].float()
        got = p.grad.float().cpu()
        assert got.shape == g32.shape
        scale = g32.abs().max().item()
        e_hip, e_ref = (got - g32).abs().max().item() / scale, (gbf - g32).abs().max().item() / scale
        worst = max(worst, e_hip)
        # HIP's distance to the exact (fp32) gradient within 1.3x the reference-bf16 run's own, or 2% of max
        assert e_hip <= max(1.3 * e_ref, 0.02), f"{n}: hip {e_hip:.3f} vs reference-bf16 {e_ref:.3f} of max|g|"
    record_parity(f"{name}.lora_grads", worst_dist_to_fp32_over_max_g=worst, loss=loss.item(), ref_loss_bf16=ref_loss, ref_loss_fp32=f32_loss)
    # eval-mode forward under no_grad still goes through the engine
    m.eval()
    with torch.no_grad():
        lg = m(idx)
    val = chunked_cross_entropy(lg[..., :-1, :], labels[..., 1:], chunk_size=0)
    assert abs(val.item() - t["fp32.val_loss"].item()) <= max(2 * abs(t["bf16.val_loss"].float().item() - t["fp32.val_loss"].item()), 5e-2)


def test_graphed_micro_step_equals_autograd():
    """GraphedTrainStep (forward + chunked CE + backward + bucket accumulation captured as one hipGraph per
    padded length) against the autograd path: same loss, same gradients; replays accumulate."""
    from dualhyp_amd import GPT, Config
    from dualhyp_amd.synth import synth_state_dict
    from dualhyp_amd.train import prepare_for_training, GraphedTrainStep
    from dualhyp_amd.finetune import FlatGradBucket, micro_loss
    cfg = Config.from_name("parity-tiny", r=4, alpha=8, dropout=0.0, to_query=True, to_key=True, to_value=True, to_projection=True)
    m = GPT(cfg).to(device=DEV, dtype=torch.bfloat16)
    m.load_state_dict(synth_state_dict(cfg, seed=5, device=DEV))
    m.train()
    params = prepare_for_training(m)
    bucket = FlatGradBucket(params)
    g = torch.Generator().manual_seed(2)
    step = GraphedTrainStep(m, bucket)
    for T in (37, 50, 37):                   # 37 and 50 share the padded length 64: one graph, two contents
        ids = torch.randint(3, cfg.padded_vocab_size, (1, T), generator=g).to(DEV)
        labels = ids.clone()
        labels[:, : T // 2] = -1
        bucket.zero()
        loss = micro_loss(m, ids, labels, 8)
        (loss / 4).backward()
        want_loss, want = loss.item(), bucket.flat.clone()
        bucket.zero()
        got_loss = step(ids, labels, 1.0 / 4).item()
        assert abs(got_loss - want_loss) <= 1e-5 * max(1.0, abs(want_loss)), (got_loss, want_loss)
        err = (bucket.flat - want).abs().max().item() / want.abs().max().item()
        assert err <= 2e-3, f"T={T}: graphed gradients differ from autograd by {err:.2e} of max|g|"
        step(ids, labels, 1.0 / 4)            # a second replay adds the same gradient again
        assert (bucket.flat - 2 * want).abs().max().item() / want.abs().max().item() <= 4e-3
    assert len(step._graphs) == 1


def test_fit_reduces_loss_and_saves_reference_checkpoint(tmp_path):
    """A few optimizer steps of the whole loop (AdamW on fp32 LoRA masters, flat grad bucket, LR warm-up,
    validation through the inference engine, checkpoint in the reference's format)."""
    from dualhyp_amd import GPT, Config
    from dualhyp_amd.data import collate
    from dualhyp_amd.finetune import TrainConfig, fit, validate
    from dualhyp_amd.synth import synth_state_dict, hash_u24, stream_id
    cfg = Config.from_name("parity-tiny", r=4, alpha=8, dropout=0.05, to_query=True, to_key=True, to_value=True, to_projection=True)
    sd = synth_state_dict(cfg, seed=5, weight_scale=4.0, device=DEV)
    m = GPT(cfg).to(device=DEV, dtype=torch.bfloat16)
    m.load_state_dict(sd)
    exs = []
    for i in range(16):            # a learnable toy task: the response repeats a fixed pattern
        T = 20 + i % 5
        ids = (hash_u24(T, stream_id(9, f"ex{i}")) % 200 + 3)
        ids[-6:] = torch.tensor([7, 8, 9, 7, 8, 2])
        lab = ids.clone()
        lab[:-6] = -1
        exs.append({"input_ids": ids, "labels": lab, "input_ids_no_response": ids[:-6], "input": "", "uid": str(i), "ground_truth": ""})
    val = lambda: [collate(exs[:4])]
    val = lambda: [{k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in collate(exs[:4]).items()}]
    v0 = validate(m, val())
    tc = TrainConfig(learning_rate=2e-3, num_epochs=6, batch_size=4, micro_batch_size=2, lm_head_chunk_size=8)
    out = fit(m, exs, collate, tc, val_batches=val, out_dir=str(tmp_path), device=DEV, log=lambda s: None)
    assert out["optimizer_steps"] == 6 * 16 // 4
    assert out["best_val_loss"] < 0.7 * v0, (v0, out)
    ck = torch.load(tmp_path / "best_model.pth")
    assert set(ck) == {"model"} and "transformer.h.1.attn.attn.lora_B" in ck["model"] and "lm_head.linear.weight" in ck["model"]
    # the LoRA masters moved, the frozen base did not
    assert not torch.equal(ck["model"]["transformer.h.0.attn.proj.lora_B"].float(), sd["transformer.h.0.attn.proj.lora_B"].float().cpu())
    assert torch.equal(ck["model"]["transformer.h.0.mlp.fc_1.linear.weight"], sd["transformer.h.0.mlp.fc_1.linear.weight"].cpu())


def test_train_micro_step_tinyllama_shape(golden):
    """BASELINE config 3 at the TinyLlama layer shape (d 2048, 32/4 heads, I 5632, V 32000, LoRA r 16; 2 layers),
    one micro-batch of T = 560 tokens (512 prompt positions masked, 47 response tokens + EOS): loss and every LoRA
    gradient of finetune/ger.py:278-285 against what the REFERENCE's autograd produced in fp32, in bf16-true and under
    its own training precision bf16-mixed (fp32 parameters + autocast).  HIP keeps activations in bf16 and LoRA
    masters in fp32: its distance to the fp32 gradient must stay within 2x that of the reference's own bf16-true and
    bf16-mixed runs (whichever is larger)."""
    from dualhyp_amd import GPT, Config, chunked_cross_entropy
    from dualhyp_amd.synth import synth_state_dict
    from dualhyp_amd.train import prepare_for_training
    t, meta = golden("train_tinyllama_shape")
    cfg = Config(**meta["config"])
    sd = synth_state_dict(cfg, seed=meta["seed"], norm_jitter=meta["norm_jitter"], device=DEV)
    m = GPT(cfg).to(device=DEV, dtype=torch.bfloat16)
    m.load_state_dict(sd)
    m.cpu_rsqrt_vec_width = 32
    m.train()
    prepare_for_training(m)
    ids, labels = t["input_ids"].to(DEV), t["labels"].to(DEV)
    logits = m(ids, lm_head_chunk_size=128)
    assert len(logits) == 5 and logits[0].shape == (1, 128, 32000)
    logits[-1] = logits[-1][..., :-1, :]
    loss = chunked_cross_entropy(logits, labels[..., 1:], chunk_size=128)
    (loss / meta["grad_accum"]).backward()
    l32, lbf, lmx = (t[f"{k}.train_loss"].float().item() for k in ("fp32", "bf16", "mixed"))
    assert abs(loss.item() - l32) <= max(1.3 * abs(lbf - l32), 1.3 * abs(lmx - l32), 1e-3), (loss.item(), l32, lbf, lmx)
    worst = {"hip": 0.0, "bf16": 0.0, "mixed": 0.0}
    for n, p in m.named_parameters():
        if "lora_" not in n:
            continue
        g32 = t[f"fp32.grad.{n}"].float()
        scale = g32.abs().max().item()
        e = {"hip": (p.grad.float().cpu() - g32).abs().max().item() / scale,
             "bf16": (t[f"bf16.grad.{n}"].float() - g32).abs().max().item() / scale,
             "mixed": (t[f"mixed.grad.{n}"].float() - g32).abs().max().item() / scale}
        for k in worst:
            worst[k] = max(worst[k], e[k])
        assert e["hip"] <= max(1.3 * max(e["bf16"], e["mixed"]), 0.02), f"{n}: {e}"
    assert worst["hip"] <= 1.3 * max(worst["bf16"], worst["mixed"]), worst      # 2x -> 1.3x (measured 1.02x)
    record_parity("train_tinyllama_shape.lora_grads", loss_hip=loss.item(), loss_fp32=l32, loss_bf16=lbf, loss_mixed=lmx,
                  worst_hip=worst["hip"], worst_ref_bf16=worst["bf16"], worst_ref_mixed=worst["mixed"])


def test_train_micro_step_full_depth(golden):
    """The same micro-step at FULL depth: all 22 layers of TinyLlama-1.1B (VERDICT r02 missing #7), T = 560, against the REFERENCE's
    autograd in fp32, bf16-true and bf16-mixed (tests/golden/train_tinyllama_full: the fp32 LoRA gradients of layers 0 and 21 in full with
    the bf16 / mixed runs' distances to them, max |g| and the norm of every other layer's fp32 gradient).  Through 22 layers of bf16 activations the backward signal
    of layer 0 has passed every kernel of the path: HIP's distance to the fp32 gradient within 1.3x the reference's own bf16 runs'."""
    from dualhyp_amd import GPT, Config, chunked_cross_entropy
    from dualhyp_amd.synth import synth_state_dict
    from dualhyp_amd.train import prepare_for_training
    t, meta = golden("train_tinyllama_full")
    cfg = Config(**meta["config"])
    assert cfg.n_layer == 22 and meta["keep_layers"] == [0, 21]
    m = GPT(cfg).to(device=DEV, dtype=torch.bfloat16)
    m.load_state_dict(synth_state_dict(cfg, seed=meta["seed"], norm_jitter=meta["norm_jitter"], device=DEV))
    m.cpu_rsqrt_vec_width = 32
    m.train()
    prepare_for_training(m)
    ids, labels = t["input_ids"].to(DEV), t["labels"].to(DEV)
    logits = m(ids, lm_head_chunk_size=128)
    logits[-1] = logits[-1][..., :-1, :]
    loss = chunked_cross_entropy(logits, labels[..., 1:], chunk_size=128)
    (loss / meta["grad_accum"]).backward()
    l32, lbf, lmx = (t[f"{k}.train_loss"].float().item() for k in ("fp32", "bf16", "mixed"))
    assert abs(loss.item() - l32) <= max(1.3 * abs(lbf - l32), 1.3 * abs(lmx - l32), 2e-3), (loss.item(), l32, lbf, lmx)
    worst = {"hip": 0.0, "bf16": 0.0, "mixed": 0.0}
    norm_ratio = []
    for n, p in m.named_parameters():
        if "lora_" not in n:
            continue
        g = p.grad.float().cpu()
        if f"fp32.grad.{n}" not in t:                         # a layer kept as statistics only
            stat = t[f"fp32.gradstat.{n}"]
            norm_ratio.append(g.norm().item() / max(stat[1].item(), 1e-30))
            continue
        g32 = t[f"fp32.grad.{n}"].float()
        scale = g32.abs().max().item()
        e = {"hip": (g - g32).abs().max().item() / scale, "bf16": t[f"bf16.graderr.{n}"].item(), "mixed": t[f"mixed.graderr.{n}"].item()}
        for k in worst:
            worst[k] = max(worst[k], e[k])
        assert e["hip"] <= max(1.3 * max(e["bf16"], e["mixed"]), 0.02), f"{n}: {e}"
    record_parity("train_tinyllama_full.lora_grads", loss_hip=loss.item(), loss_fp32=l32, loss_bf16=lbf, loss_mixed=lmx, worst_hip=worst["hip"],
                  worst_ref_bf16=worst["bf16"], worst_ref_mixed=worst["mixed"], other_layers_norm_ratio_min=min(norm_ratio), other_layers_norm_ratio_max=max(norm_ratio))
    assert worst["hip"] <= 1.3 * max(worst["bf16"], worst["mixed"]), worst
    assert 0.9 <= min(norm_ratio) and max(norm_ratio) <= 1.1, (min(norm_ratio), max(norm_ratio))


def test_adamw_trajectory_matches_reference(golden):
    """Three optimizer steps of the fine-tune loop (AdamW lr/weight-decay, warm-up schedule, accumulation over 2
    micro-batches of one utterance; finetune/ger.py:126-133,255-292) through dualhyp_amd.finetune.fit against the
    trajectory produced around the REFERENCE's model (tests/golden/adamw_tiny: fp32 and bf16-mixed).  Adam's first
    steps move every element by ~lr whatever the gradient's size, so the yardstick is the distance between the
    reference's own two precisions, relative to the size of the update."""
    from dualhyp_amd import GPT, Config
    from dualhyp_amd.data import collate
    from dualhyp_amd.finetune import TrainConfig, fit
    from dualhyp_amd.synth import synth_state_dict
    t, meta = golden("adamw_tiny")
    cfg = Config(**meta["config"])
    sd = synth_state_dict(cfg, seed=meta["seed"], norm_jitter=meta["norm_jitter"], weight_scale=meta["weight_scale"], device=DEV)
    m = GPT(cfg).to(device=DEV, dtype=torch.bfloat16)
    m.load_state_dict(sd)
    m.cpu_rsqrt_vec_width = 32
    n = meta["accum"] * meta["steps"]
    exs = []
    for i in range(n):
        ids = t[f"ids{i}"]
        lab = ids.clone()
        lab[: ids.numel() - 9] = -1
        exs.append({"input_ids": ids, "labels": lab, "input_ids_no_response": ids[:-9], "input": "", "uid": str(i), "ground_truth": ""})
    # warm-up = int(epoch_size * warmup_frac) // world = int(6 * 0.7) = 4 iterations, as the fixture's loop
    tc = TrainConfig(learning_rate=meta["lr"], weight_decay=meta["weight_decay"], num_epochs=1, batch_size=meta["accum"],
                     micro_batch_size=1, warmup_frac=0.7, lm_head_chunk_size=8, shuffle=False)
    snaps, losses = [], []
    names = [k for k, _ in m.named_parameters() if "lora_" in k]
    fit(m, exs, collate, tc, device=DEV, log=lambda s: None,
        on_step=lambda step, ps: snaps.append({k: p.detach().float().cpu().clone() for k, p in m.named_parameters() if "lora_" in k}),
        on_micro=lambda it, loss: losses.append(loss.float().item()))
    assert len(snaps) == meta["steps"] and len(losses) == n
    l32, lmx = t["fp32.losses"], t["mixed.losses"]
    for i, l in enumerate(losses):
        assert abs(l - l32[i].item()) <= max(3 * abs(lmx[i].item() - l32[i].item()), 3e-2), (i, l, l32[i].item(), lmx[i].item())
    base = {k: v.float().cpu() for k, v in sd.items() if "lora_" in k}
    for step, snap in enumerate(snaps):
        num_h = num_m = den = 0.0
        for k in names:
            a, mx = t[f"fp32.step{step}.{k}"], t[f"mixed.step{step}.{k}"]
            num_h += (snap[k] - a).pow(2).sum().item()
            num_m += (mx - a).pow(2).sum().item()
            den += (a - base[k]).pow(2).sum().item()
        rh, rm = (num_h / den) ** 0.5, (num_m / den) ** 0.5
        record_parity(f"adamw_tiny.step{step}", rel_rms_hip_vs_fp32=rh, rel_rms_mixed_vs_fp32=rm, loss_hip=losses[2 * step + 1],
                      loss_fp32=l32[2 * step + 1].item())
        assert rh <= max(1.3 * rm, 0.02), f"step {step}: HIP parameters {rh:.3f} of the update away from the fp32 trajectory, bf16-mixed {rm:.3f}"


def test_noise_mask_classifier_training():
    """SURVEY §8f-3, training half (finetune/relprompt.py:356-387): the mask cross entropy and the gradients of all six
    NoiseMaskClassifier parameters on the HIP path against what the REFERENCE module's autograd produced
    (tests/golden/noise_mask_classifier: ger/relprompt.py:126-147 in fp32, in bf16 and under bf16 autocast with fp32
    parameters).  HIP keeps activations in bf16 and accumulates parameter gradients in fp32: per parameter its distance
    to the fp32 gradient must stay within 1.3x the larger of the reference's own bf16 / bf16-mixed distances."""
    import torch.nn.functional as F
    from conftest import load_golden
    from dualhyp_amd.relprompt import NoiseMaskClassifier
    t, meta = load_golden("noise_mask_classifier")
    for tag, mt in meta.items():
        C, pool, T = mt["C"], mt["pool"], mt["T"]
        m = NoiseMaskClassifier(C, pool_size=pool).eval()             # eval: the fixture has no dropout draw
        sd = {k: uniform(tuple(shape), 1.0 / math.sqrt(math.prod(shape[1:]) if len(shape) > 1 else 256.0), stream_id(mt["seed"], tag + k)).float()
              for k, shape in mt["shapes"].items()}
        m.load_state_dict(sd)
        m = m.to(DEV)
        x = uniform((2, T, C), 1.5, stream_id(mt["seed"], tag + "x")).to(DEV)
        logits = m(x)
        assert logits.requires_grad and logits.shape == (2, (T + pool - 1) // pool, 3)
        with torch.no_grad():
            assert torch.equal(logits.detach(), m(x)), "the differentiable forward must produce the inference path's bits"
        loss = F.cross_entropy(logits.float().view(-1, 3), t[f"{tag}.targets"].to(DEV).view(-1))
        loss.backward()
        l32, lbf, lmx = (t[f"{tag}.{k}.loss"].item() for k in ("fp32", "bf16", "mixed"))
        assert abs(loss.item() - l32) <= max(1.3 * abs(lbf - l32), 1.3 * abs(lmx - l32), 1e-3), (loss.item(), l32, lbf, lmx)
        worst = {}
        for k, p in m.named_parameters():
            g = p.grad.float().cpu()
            assert g.shape == p.shape
            if g.dim() == 3:
                stat = t[f"{tag}.fp32.gradstat.{k}"]
                assert abs(g.norm().item() - stat[1].item()) <= 0.05 * stat[1].item(), (k, g.norm().item(), stat[1].item())
                scale, g = stat[0].item(), g[::16]
            else:
                scale = t[f"{tag}.fp32.grad.{k}"].abs().max().item()
            g32 = t[f"{tag}.fp32.grad.{k}"]
            e = {"hip": (g - g32).abs().max().item() / scale,
                 "bf16": (t[f"{tag}.bf16.grad.{k}"] - g32).abs().max().item() / scale,
                 "mixed": (t[f"{tag}.mixed.grad.{k}"] - g32).abs().max().item() / scale}
            worst[k] = e
            assert e["hip"] <= max(1.3 * max(e["bf16"], e["mixed"]), 5e-3), f"{tag}.{k}: {e}"
        record_parity(f"classifier_training.{tag}", loss_hip=loss.item(), loss_fp32=l32, loss_bf16=lbf, loss_mixed=lmx,
                      **{f"{k}.{w}": v for k, e in worst.items() for w, v in e.items()})


def test_fit_trains_relprompt_classifiers():
    """finetune/relprompt.py:175-195,356-403 through `fit`: the reliability classifiers are the optimizer's second
    group with their own learning rate, trained on 0.02 x (audio + visual mask cross entropy) added to the LM loss."""
    from dualhyp_amd import Config
    from dualhyp_amd.data import collate
    from dualhyp_amd.finetune import TrainConfig, fit, micro_loss
    from dualhyp_amd.relprompt import GPT as RelGPT, mask_loss
    from dualhyp_amd.synth import synth_state_dict, hash_u24, stream_id
    cfg = Config.from_name("parity-tiny", r=4, alpha=8, dropout=0.0, to_query=True, to_key=True, to_value=True, to_projection=True)
    cfg.whisper_dim, cfg.raven_dim, cfg.pool_size = 64, 32, 2

    def fresh():
        torch.manual_seed(3)
        m = RelGPT(cfg)
        m.load_state_dict(synth_state_dict(cfg, seed=5, weight_scale=4.0), strict=False)
        m.resize_token_embeddings(3)
        return m.to(device=DEV, dtype=torch.bfloat16)

    exs = []
    for i in range(8):
        T = 20 + i % 3
        ids = (hash_u24(T, stream_id(9, f"rp{i}")) % 200 + 3)
        lab = ids.clone()
        lab[:-6] = -1
        Ta, Tv = 16, 8                                      # 50 fps / 25 fps features; pools 4 and 2 -> 4 chunks each
        exs.append({"input_ids": ids, "labels": lab,
                    "audio_enc_features": uniform((Ta, 64), 1.0, stream_id(9, f"a{i}")).float(),
                    "visual_enc_features": uniform((Tv, 32), 1.0, stream_id(9, f"v{i}")).float(),
                    "audio_mask_targets": torch.tensor([i % 3, 0, 1, 2]), "visual_mask_targets": torch.tensor([2, (i + 1) % 3, 0, 1, 1])})
    names = ["audio_noise_classifier.conv1.weight", "visual_noise_classifier.classifier.bias", "audio_noise_classifier.conv2.bias"]
    m = fresh()
    before = {k: v.detach().float().cpu().clone() for k, v in m.named_parameters() if k in names}
    # the loss fit optimises = LM loss + 0.02 x mask loss (first micro-batch, evaluated here the same way)
    seen = []
    tc = TrainConfig(learning_rate=1e-3, classifier_learning_rate=5e-3, num_epochs=1, batch_size=4, micro_batch_size=1,
                     lm_head_chunk_size=8, shuffle=False, warmup_frac=0.2)
    out = fit(m, exs, collate, tc, device=DEV, log=lambda s: None, on_micro=lambda it, loss: seen.append(loss.float().item()))
    assert out["optimizer_steps"] == 2
    after = {k: v.detach().float().cpu() for k, v in m.named_parameters() if k in names}
    assert all(p.dtype == torch.float32 and p.requires_grad for n, p in m.named_parameters() if "noise_classifier" in n)
    assert all(not torch.equal(before[k], after[k]) for k in names), "every classifier parameter must move"
    # classifier learning rate 0: the second group is really a separate group — LoRA moves, the classifiers do not
    m0 = fresh()
    lora0 = m0.transformer.h[0].attn.proj.lora_B.detach().float().cpu().clone()
    tc0 = TrainConfig(learning_rate=1e-3, classifier_learning_rate=0.0, weight_decay=0.0, num_epochs=1, batch_size=4, micro_batch_size=1,
                      lm_head_chunk_size=8, shuffle=False)
    fit(m0, exs, collate, tc0, device=DEV, log=lambda s: None)
    after0 = {k: v.detach().float().cpu() for k, v in m0.named_parameters() if k in names}
    assert all(torch.equal(before[k], after0[k]) for k in names)
    assert not torch.equal(lora0, m0.transformer.h[0].attn.proj.lora_B.detach().float().cpu())
    # the first micro-batch's loss is LM + 0.02 x (audio + visual CE)
    m1 = fresh()
    from dualhyp_amd.train import prepare_for_training
    from dualhyp_amd.relprompt import prepare_classifiers_for_training
    m1.train(); prepare_for_training(m1); prepare_classifiers_for_training(m1)
    b = collate(exs[:1])
    lm = micro_loss(m1, b["input_ids"].to(DEV), b["labels"].to(DEV), 8)
    ml = mask_loss(m1.audio_noise_classifier(b["audio_enc_features"].to(DEV)), m1.visual_noise_classifier(b["visual_enc_features"].to(DEV)),
                   b["audio_mask_targets"].to(DEV), b["visual_mask_targets"].to(DEV))
    assert ml.item() > 0.5 and abs(seen[0] - (lm.item() + 0.02 * ml.item())) <= 2e-3 * abs(seen[0]), (seen[0], lm.item(), ml.item())


def test_packed_micro_steps_equal_the_sequential_ones():
    """P micro-batches of an accumulation window as ONE packed forward / backward (GraphedTrainStep with
    [P, T] inputs) keep `micro_batch_size 1` semantics: every sequence's loss is the loss of its own micro-step and what
    lands in the bucket is the sum of the P sequential micro-steps' gradients up to fp32 round-off — ragged lengths
    (right padding inside the pack), a tiny and a d-2048 shape (where the pack moves the GEMMs onto the 256-tile kernels)."""
    from dualhyp_amd import GPT, Config
    from dualhyp_amd.synth import synth_state_dict
    from dualhyp_amd.train import prepare_for_training, GraphedTrainStep
    from dualhyp_amd.finetune import FlatGradBucket
    for name, kw, lens in (("parity-tiny", dict(r=4, alpha=8), (37, 50, 64, 23)),
                           ("parity-block", dict(r=16, alpha=16), (560, 531, 560, 498, 560, 512, 547, 560))):
        cfg = Config.from_name(name, dropout=0.0, to_query=True, to_key=True, to_value=True, to_projection=True, **kw)
        m = GPT(cfg).to(device=DEV, dtype=torch.bfloat16)
        m.load_state_dict(synth_state_dict(cfg, seed=5, device=DEV, norm_jitter=0.25))
        m.train()
        params = prepare_for_training(m)
        bucket = FlatGradBucket(params)
        step = GraphedTrainStep(m, bucket)
        g = torch.Generator().manual_seed(4)
        P, Tm = len(lens), max(lens)
        ids = torch.zeros((P, Tm), dtype=torch.int64)
        labels = torch.full((P, Tm), -1, dtype=torch.int64)
        for i, n in enumerate(lens):
            ids[i, :n] = torch.randint(3, cfg.padded_vocab_size, (n,), generator=g)
            labels[i, n // 2:n] = ids[i, n // 2:n]
        ids, labels = ids.to(DEV), labels.to(DEV)
        bucket.zero()
        seq_losses = [step(ids[i:i + 1, :n].contiguous(), labels[i:i + 1, :n].contiguous(), 1.0 / P).item() for i, n in enumerate(lens)]
        want = bucket.flat.clone()
        bucket.zero()
        got_losses = step(ids, labels, 1.0 / P, lengths=lens).tolist()
        err = (bucket.flat - want).abs().max().item() / want.abs().max().item()
        dl = max(abs(a - b) / max(1.0, abs(b)) for a, b in zip(got_losses, seq_losses))
        record_parity(f"packed_micro_steps.{name}", sequences=P, max_rel_loss_diff=dl, grad_diff_over_max_g=err)
        assert dl <= 1e-6, (got_losses, seq_losses)
        assert err <= 1e-5, f"{name}: packed gradients differ from the sum of the sequential ones by {err:.2e} of max|g|"


def test_fit_with_packing_matches_fit_without():
    """`fit(pack=4)` against `fit(pack=2)`: the same optimizer trajectory (both through GraphedTrainStep; the packs only
    change how many micro-batches share a launch), per-micro-batch losses reported in order, and against the
    launch-by-launch autograd schedule (`pack=1`) to the agreement of the two CE paths."""
    from dualhyp_amd import GPT, Config
    from dualhyp_amd.data import collate
    from dualhyp_amd.finetune import TrainConfig, fit
    from dualhyp_amd.synth import synth_state_dict, hash_u24, stream_id
    cfg = Config.from_name("parity-tiny", r=4, alpha=8, dropout=0.0, to_query=True, to_key=True, to_value=True, to_projection=True)
    exs = []
    for i in range(16):
        T = 20 + (i * 7) % 11
        ids = (hash_u24(T, stream_id(9, f"pk{i}")) % 200 + 3)
        lab = ids.clone()
        lab[:-6] = -1
        exs.append({"input_ids": ids, "labels": lab})

    def run(pack):
        m = GPT(cfg).to(device=DEV, dtype=torch.bfloat16)
        m.load_state_dict(synth_state_dict(cfg, seed=5, weight_scale=4.0, device=DEV))
        losses, snaps = {}, []
        tc = TrainConfig(learning_rate=2e-3, num_epochs=1, batch_size=8, micro_batch_size=1, lm_head_chunk_size=8, shuffle=False, pack=pack)
        out = fit(m, exs, collate, tc, device=DEV, log=lambda s: None, on_micro=lambda it, l: losses.__setitem__(it, l.float().item()),
                  on_step=lambda k, ps: snaps.append(torch.cat([p.detach().float().flatten() for p in ps]).cpu()))
        assert out["optimizer_steps"] == 2 and sorted(losses) == list(range(16))
        return [losses[i] for i in range(16)], snaps
    l4, s4 = run(4)
    l2, s2 = run(2)
    l1, s1 = run(1)
    assert max(abs(a - b) for a, b in zip(l4[:8], l2[:8])) <= 1e-6          # first window: identical parameters
    for a, b in zip(s4, s2):
        assert (a - b).abs().max().item() <= 1e-6 * max(1.0, b.abs().max().item())
    # launch by launch the loss is torch's cross entropy on bf16 logit chunks (bf16 log-softmax), packed it is the fp32 HIP
    # kernel: they agree to bf16 resolution of a loss of 2..5, and so do the trajectories (AdamW's first steps are
    # sign-like: compare in the RMS sense, against the size of the update itself)
    assert max(abs(a - b) for a, b in zip(l4[:8], l1[:8])) <= 1e-2
    base = torch.cat([v.float().flatten() for k, v in synth_state_dict(cfg, seed=5, weight_scale=4.0).items() if "lora_" in k])
    upd = (s1[1] - base).pow(2).mean().sqrt().item()
    drift = (s4[1] - s1[1]).pow(2).mean().sqrt().item()
    record_parity("fit_packing.tiny", rms_update=upd, rms_packed_vs_launch_by_launch=drift, max_loss_diff_first_window=max(abs(a - b) for a, b in zip(l4[:8], l1[:8])))
    assert drift <= 0.25 * upd, f"packed and launch-by-launch trajectories drift apart: {drift:.3e} vs an update of {upd:.3e}"


def test_graph_cache_is_bounded_and_shares_one_pool():
    """ragged data make tens of (padded length, head rows) graph keys.  All graphs capture into ONE memory
    pool and the cache is an LRU of `max_graphs` entries: device memory stays bounded however many keys arrive, an evicted key
    is captured again and gives the bits it gave the first time, and an undercounted `n_targets` is refused at capture."""
    from dualhyp_amd import GPT, Config
    from dualhyp_amd.synth import synth_state_dict
    from dualhyp_amd.train import prepare_for_training, GraphedTrainStep
    from dualhyp_amd.finetune import FlatGradBucket
    cfg = Config.from_name("parity-block", r=16, alpha=16, dropout=0.0, to_query=True, to_key=True, to_value=True, to_projection=True)
    m = GPT(cfg).to(device=DEV, dtype=torch.bfloat16)
    m.load_state_dict(synth_state_dict(cfg, seed=5, device=DEV, norm_jitter=0.25))
    m.train()
    params = prepare_for_training(m)
    bucket = FlatGradBucket(params)
    step = GraphedTrainStep(m, bucket, max_graphs=3)
    g = torch.Generator().manual_seed(8)
    P = 4

    def batch(T, n_resp):
        ids = torch.randint(3, cfg.padded_vocab_size, (P, T), generator=g)
        labels = ids.clone()
        labels[:, : T - n_resp] = -1
        return ids.to(DEV), labels.to(DEV)

    keys = [(64 * k, r) for k in range(2, 9) for r in (20, 90)]          # 14 distinct (T_pad, n_rows) keys
    first, reserved = {}, []
    for T, r in keys:
        ids, labels = batch(T, r)
        bucket.zero()
        loss = step(ids, labels, 0.25, n_targets=P * r)
        first[(T, r)] = (ids, labels, loss.clone(), bucket.flat.clone())
        torch.cuda.synchronize()
        reserved.append(torch.cuda.memory_reserved())
        assert len(step._graphs) <= 3
    assert step.captures == len(keys)
    # after the LARGEST key of the first three the pool only grows by what a larger graph needs: the 14 graphs do not add up
    act = lambda T: P * T * 2 * cfg.n_embd * 2       # a loose lower bound of one graph's saved activations (bytes)
    assert reserved[-1] - reserved[2] < 0.5 * sum(act(T) for T, _ in keys[3:]) + (256 << 20), reserved
    # an evicted key comes back: captured again, same bits
    T, r = keys[0]
    ids, labels, loss0, flat0 = first[keys[0]]
    assert (P, T, m.cpu_rsqrt_vec_width, 256) not in step._graphs
    bucket.zero()
    loss = step(ids, labels, 0.25, n_targets=P * r)
    assert step.captures == len(keys) + 1 and torch.equal(loss, loss0) and torch.equal(bucket.flat, flat0)
    # an undercount is refused where the key is captured (512 targets, 256 head rows)
    ids, labels = batch(192, 128)
    with pytest.raises(ValueError, match="n_targets"):
        step(ids, labels, 0.25, n_targets=200)


def test_lora_dropout_kernel():
    """dh_dropout_bf16 (ABI 5): the LoRA-branch dropout of ger/lora.py:96,165,391 with the mask drawn in the kernel.  The values are
    torch's (x * (keep.to(bf16) * 1/(1-p)) in bf16), the keep rate is 1 - p, the draws are a pure function of (seed, call id,
    device step counter, element) — and a captured hipGraph that bumps the counter draws a fresh mask on every replay."""
    from dualhyp_amd import ops
    p, n = 0.05, 1 << 20
    x = torch.randn(n // 2048, 2048, device=DEV).bfloat16()
    step = torch.zeros(1, dtype=torch.int64, device=DEV)
    y, m = ops.dropout(x, p, seed=1337, call_id=3, step=step)
    keep = m != 0
    scale = torch.tensor(1.0 / (1.0 - p)).bfloat16()
    assert torch.equal(m[keep], scale.to(DEV).expand_as(m[keep])) and float(m[~keep].abs().sum()) == 0
    assert torch.equal(y, x * m)                                    # torch's bf16 multiply: the same rounding
    rate = keep.float().mean().item()
    assert abs(rate - (1 - p)) < 4 * (p * (1 - p) / n) ** 0.5 + 1e-4, rate
    # no structure along rows / columns / 8-element groups (a broken counter would repeat or stripe)
    assert abs(keep.float().mean(0) - (1 - p)).max().item() < 0.06 and abs(keep.float().mean(1) - (1 - p)).max().item() < 0.03
    k8 = keep.view(-1, 8).float()
    assert abs(torch.corrcoef(k8.T)[0, 1:]).max().item() < 0.01
    # a pure function of its key and counter
    y2, m2 = ops.dropout(x, p, seed=1337, call_id=3, step=step)
    assert torch.equal(m2, m) and torch.equal(y2, y)
    for kw in (dict(seed=1338, call_id=3), dict(seed=1337, call_id=4)):
        _, mo = ops.dropout(x, p, step=step, **kw)
        assert 0.85 < (mo == m).float().mean().item() < 0.95        # independent masks agree on (1-p)^2 + p^2 = 0.905
    step.add_(1)
    _, m3 = ops.dropout(x, p, seed=1337, call_id=3, step=step)
    assert 0.85 < (m3 == m).float().mean().item() < 0.95
    # under a hipGraph: the counter bump is captured, every replay draws anew
    g, st = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.stream(st):
        step.add_(1); ops.dropout(x, p, seed=1337, call_id=3, step=step)       # warm-up outside capture
        torch.cuda.current_stream().synchronize()
        with torch.cuda.graph(g, stream=st):
            step.add_(1)
            _, mg = ops.dropout(x, p, seed=1337, call_id=3, step=step)
    g.replay(); torch.cuda.synchronize(); a = mg.clone()
    g.replay(); torch.cuda.synchronize(); b = mg.clone()
    assert 0.85 < (a == b).float().mean().item() < 0.95
    # p = 0 keeps everything
    y0, m0 = ops.dropout(x, 0.0, seed=1, call_id=0, step=step)
    assert torch.equal(y0, x) and bool((m0 == 1).all())


@pytest.mark.parametrize("M,I,K", [(1000, 5632, 512), (1024, 5600, 256), (40, 384, 256)])
def test_swiglu_train_forward_is_the_three_launch_form(M, I, K):
    """dh_linear_swiglu_train_bf16 (one GEMM launch that also stores fc_1(x), fc_2(x)) against linear, linear, swiglu_fwd: the same
    bits, on full tiles, on a ragged M and column edge, and below the 256-tile kernel's range (where it IS the three launches)."""
    from dualhyp_amd import ops
    x, w1, w2 = U((M, K), 1.0, f"stx{M}").to(DEV), U((I, K), 0.08, f"stw1{I}").to(DEV), U((I, K), 0.08, f"stw2{I}").to(DEV)
    act, g, u = ops.linear_swiglu_train(x, w1, w2)
    g0, u0 = ops.linear(x, w1), ops.linear(x, w2)
    assert torch.equal(g, g0) and torch.equal(u, u0)
    assert torch.equal(act, ops.swiglu_fwd(g0, u0))
    assert torch.equal(act, ops.linear(x, w1, epilogue=ops.EPI_SWIGLU, w2=w2))


@pytest.mark.parametrize("hs,heads", [(64, 5), (128, 2)])
def test_fragment_order_transpose_from_the_inverse_map(hs, heads):
    """dh_transpose_frag_bf16 (16-byte accesses, inverse map, padding written as zeros) against dh_transpose_pad_bf16 into a zeroed
    destination: the same bytes, sequences that end inside a 32-token tile included."""
    from dualhyp_amd import ops, _lib
    lib = _lib.load()
    lens = [70, 33, 1, 128, 31]
    n_tok = sum(lens)
    i32 = torch.int32
    src = U((n_tok, heads, hs), 1.0, f"tf{hs}").to(DEV)
    starts = torch.tensor([sum(lens[:i]) for i in range(len(lens))], dtype=i32, device=DEV)
    qlen = torch.tensor(lens, dtype=i32, device=DEV)
    plan = ops.attn_bwd_plan(starts, qlen, n_tok, lens)
    n_pad = plan["n_pad"]
    assert n_pad == sum(-(-n // 32) * 32 for n in lens) and int((plan["pad_tok"] >= 0).sum()) == n_tok
    tok_seq = torch.repeat_interleave(torch.arange(len(lens), dtype=i32, device=DEV), qlen.long())
    want = torch.zeros((heads, hs, n_pad), dtype=torch.bfloat16, device=DEV)
    ops.check(lib.dh_transpose_pad_bf16(src.data_ptr(), want.data_ptr(), tok_seq.data_ptr(), starts.data_ptr(), plan["pad_start"].data_ptr(),
                                        n_tok, heads, hs, n_pad, torch.cuda.current_stream().cuda_stream))
    got = torch.full((heads, hs, n_pad), 7.0, dtype=torch.bfloat16, device=DEV)
    ops.check(lib.dh_transpose_frag_bf16(src.data_ptr(), got.data_ptr(), plan["pad_tok"].data_ptr(), heads, hs, n_pad,
                                         torch.cuda.current_stream().cuda_stream))
    assert torch.equal(got, want)


def test_row_split_of_a_grid_with_a_small_last_round_keeps_the_bits():
    """dh_linear_bf16 on 17 920 x 2048 (560 tiles of 256 x 256 = 2.19 rounds of the chip): the rows of the two full rounds on the
    256-tile kernel + the last 1 536 rows on the 128-tile kernel against the single launch (dh_set_tuning(28, 0)); plain, with a
    residual, and with the LoRA epilogue (xa from memory)."""
    from dualhyp_amd import ops, _lib
    lib = _lib.load()
    M, N, K = 32 * 560, 2048, 256
    x, w, r = U((M, K), 1.0, "rsx").to(DEV), U((N, K), 0.08, "rsw").to(DEV), U((M, N), 1.0, "rsr").to(DEV)
    xa, lb = U((M, 16), 1.0, "rsxa").to(DEV), U((N, 16), 0.1, "rslb").to(DEV)
    cases = [dict(), dict(resid=r), dict(epilogue=ops.EPI_LORA, xa=xa, lora_b=lb, lora_scale=2.0, resid=r)]
    try:
        got = [ops.linear(x, w, **kw) for kw in cases]
        lib.dh_set_tuning(28, 0)
        want = [ops.linear(x, w, **kw) for kw in cases]
    finally:
        lib.dh_set_tuning(28, 1)
    for g_, w_ in zip(got, want):
        assert torch.equal(g_, w_)


@pytest.mark.parametrize("T", [300, 4480])
def test_segmented_token_contraction_is_the_three_calls(T):
    """dh_tn_accum_seg_f32 (the three LoRA-B gradients of a fused QKV projection in one launch) against one dh_tn_accum_f32 per
    segment: the same kernel and chains, the same bits; and torch fp32."""
    from dualhyp_amd import ops
    qd, s0, s1 = 640, 384, 512
    a, b = U((T, qd), 1.0, f"sga{T}").to(DEV), U((T, 48), 1.0, f"sgb{T}").to(DEV)
    got = torch.empty((qd, 16), dtype=torch.float32, device=DEV)
    ops.tn_accum(a, b, got, scale=0.5, accumulate=False, splits=(s0, s1))
    want = torch.empty_like(got)
    bounds = (0, s0, s1, qd)
    for seg in range(3):
        ops.tn_accum(a[:, bounds[seg]:bounds[seg + 1]], b[:, 16 * seg:16 * seg + 16], want[bounds[seg]:bounds[seg + 1]], scale=0.5, accumulate=False)
    assert torch.equal(got, want)
    ref = torch.cat([0.5 * a[:, bounds[i]:bounds[i + 1]].float().T @ b[:, 16 * i:16 * i + 16].float() for i in range(3)])
    assert (got - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() + 1e-4


@pytest.mark.parametrize("M,N,K", [(17920, 2048, 64), (4200, 2000, 64), (4200, 2048, 256), (300, 512, 64)])
def test_gemm_with_a_multiplier_in_the_epilogue(M, N, K):
    """dh_linear_mul_bf16 (the LoRA branch's dropout mask applied where the backward's rank-r product is rounded) against
    dh_linear_bf16 followed by a bf16 multiply: the same bits — full tiles, ragged rows and a ragged column tile, and the small-shape
    fallback of ops.linear_mul."""
    from dualhyp_amd import ops
    x, w = U((M, K), 1.0, f"lmx{M}").to(DEV), U((N, K), 0.2, f"lmw{N}").to(DEV)
    keep = (U((M, N), 1.0, f"lmm{M}").float() > -0.9).to(torch.bfloat16).to(DEV)
    mask = (keep * (1.0 / 0.95)).to(torch.bfloat16)
    got = ops.linear_mul(x, w, mask)
    assert torch.equal(got, ops.linear(x, w) * mask)
    # K = 64 runs on gemm_k64_kernel (with and without the multiplier): the bits of the tiled kernels (dh_set_tuning(31, 0))
    from dualhyp_amd import _lib
    try:
        _lib.load().dh_set_tuning(31, 0)
        want = ops.linear(x, w) * mask
        want_mul = ops.linear_mul(x, w, mask)
    finally:
        _lib.load().dh_set_tuning(31, 1)
    assert torch.equal(got, want) and torch.equal(got, want_mul)
