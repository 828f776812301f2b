"""GPU parity of the whole HIP decoder (dualhyp_amd.GPT -> native engine) against tensors the
REFERENCE produced (tests/golden) through the reference's own API surface:
GPT.forward(idx[, input_pos]), generate().

Tolerance.  north_star asks for "logits within 1e-3 bf16".  Measured on the reference itself
(tests/golden, CPU): its bf16 run is 1.8e-2..2.3e-2 (relative RMS) away from its own fp32 run
of the same weights, and its cache and no-cache paths differ from each other by 1.8e-2..2.6e-2;
a 1-ulp difference anywhere is amplified by the next GEMM into 1-ulp flips of ~10%% of its
outputs.  An absolute 1e-3 on bf16 logits of magnitude ~1-5 (1 ulp = 4e-3..3e-2) is therefore
not a property the reference has with respect to itself.  The gates below are the strongest
ones that ARE meaningful for bf16 storage:
  * every kernel, fed the reference's own inputs, reproduces the reference's output bit for bit
    on >= 98%% of elements and within 1 bf16 ulp elsewhere (test_block_intermediates, test_hip_ops);
  * end to end, HIP-vs-reference relative RMS <= 1.0 x (reference-bf16 vs reference-fp32), i.e.
    the HIP logits are closer to the reference's than the reference's are to the exact answer,
    and HIP's max distance to the fp32 run <= 1.5 x the reference-bf16 run's;
  * greedy token ids identical wherever the reference's top-2 margin is >= 4 bf16 ulps
    (below that the reference's own arg-max is inside its bf16 noise).
"""
import os

import pytest
import torch

from conftest import ulp_diff, record_parity
from dualhyp_amd import GPT, Config, generate, generate_batch, merge_lora_weights
from dualhyp_amd.synth import synth_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TINY = ["tiny_r4", "tiny_hs128_r16"]


def build(meta, cls=GPT, extra=None, **kw):
    cfg = Config(**meta["config"])
    sd = synth_state_dict(cfg, seed=meta["seed"], norm_jitter=meta.get("norm_jitter", 0.0),
                          weight_scale=meta.get("weight_scale", 1.0), head_peak=meta.get("head_peak", 0.0),
                          embed_scale=meta.get("embed_scale", 1.0), head_tie=meta.get("head_tie", 0.0), device=DEV)
    m = cls(cfg).to(device=DEV, dtype=torch.bfloat16)
    m.load_state_dict(sd, strict=extra is None)
    m.eval()
    # the goldens are CPU tensors of the reference: reproduce torch's CPU bf16 rsqrt tail rounding (Q11); the
    # product default is 0 (single rounding, what a GPU run of the reference does)
    m.cpu_rsqrt_vec_width = 32
    return cfg, m


def rel_rms(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()


SAFE_MARGIN_ULPS = 4


def _equal_prefix(a, b) -> int:
    ne = (a != b).nonzero().flatten()
    return int(ne[0]) if ne.numel() else int(a.numel())


def gate(got, ref_bf16, ref_fp32, what, frac=1.0, abs_frac=1.5):
    got = got.float().cpu()
    e_hip = (got - ref_fp32.float()).abs().max().item()
    e_ref = (ref_bf16.float() - ref_fp32.float()).abs().max().item()
    rr = rel_rms(got, ref_bf16)
    yard = rel_rms(ref_bf16, ref_fp32)
    exact = (got == ref_bf16.float()).float().mean().item()
    record_parity(what, rel_rms_hip_vs_ref=rr, rel_rms_ref_vs_fp32=yard, bit_exact_frac=exact, max_abs_hip_vs_fp32=e_hip,
                  max_abs_ref_vs_fp32=e_ref)
    assert rr <= frac * yard, f"{what}: relative RMS {rr:.3e} > {frac} x {yard:.3e}"
    assert e_hip <= abs_frac * e_ref + 1e-3, f"{what}: HIP is {e_hip:.3e} from fp32 truth, reference bf16 only {e_ref:.3e}"


@pytest.mark.parametrize("name", TINY)
def test_forward_nocache_and_cache(golden, name):
    t, meta = golden(name)
    cfg, m = build(meta)
    T = meta["T"]
    idx = torch.stack([t["idx0"], t["idx1"]]).to(DEV)
    with torch.no_grad():
        lg = m(idx)
        gate(lg, t["bf16.logits_nocache"], t["fp32.logits_nocache"], f"{name} no-cache logits")
        chunks = m(idx, lm_head_chunk_size=8)
        assert isinstance(chunks, list) and torch.equal(torch.cat(chunks, 1), lg)
        m.reset_cache()
        lp = m(t["idx0"].view(1, -1).to(DEV), torch.arange(T, device=DEV))
        gate(lp, t["bf16.logits_prefill"], t["fp32.logits_prefill"], f"{name} prefill logits")
        same_path = t["fp32.decode_tokens"].tolist() == t["bf16.decode_tokens"].tolist()
        for s, tok in enumerate(t["bf16.decode_tokens"].tolist()):
            ld = m(torch.tensor([[tok]], device=DEV), torch.tensor([T + s], device=DEV))
            if same_path:
                gate(ld[0, 0], t["bf16.logits_decode"][s], t["fp32.logits_decode"][s], f"{name} decode step {s}", frac=1.0)
            else:
                assert rel_rms(ld[0, 0], t["bf16.logits_decode"][s]) < 3e-2


def test_cached_forward_with_grad_enabled(golden):
    """`model(x, input_pos)` OUTSIDE torch.no_grad() — the reference allows it, and requires_grad=True is the
    nn.Module default that .eval() does not change — is incremental decoding through the KV cache, not a training
    forward from position 0; a model without LoRA (r = 0) with grad enabled is plain inference."""
    t, meta = golden("tiny_r4")
    cfg, m = build(meta)
    assert any(p.requires_grad for p in m.parameters())
    T = meta["T"]
    x = t["idx0"].view(1, -1).to(DEV)
    tok = torch.tensor([[int(t["bf16.decode_tokens"][0])]], device=DEV)
    with torch.no_grad():
        want_p = m(x, torch.arange(T, device=DEV))
        want_d = m(tok, torch.tensor([T], device=DEV))
        m.reset_cache()
    got_p = m(x, torch.arange(T, device=DEV))                  # grad mode on
    got_d = m(tok, torch.tensor([T], device=DEV))
    assert torch.equal(got_p, want_p) and torch.equal(got_d, want_d) and not got_d.requires_grad
    m.reset_cache()
    cfg0 = Config(**{**meta["config"], "r": 0})
    sd0 = {k: v for k, v in synth_state_dict(cfg0, seed=meta["seed"], norm_jitter=0.25, weight_scale=4.0, device=DEV).items()}
    m0 = GPT(cfg0).to(device=DEV, dtype=torch.bfloat16)
    m0.load_state_dict(sd0, strict=True)
    m0.eval()
    lg = m0(x)                                                  # grad enabled, nothing trainable: inference path
    with torch.no_grad():
        assert torch.equal(lg, m0(x))


@pytest.mark.parametrize("name", TINY)
def test_generate_ids(golden, name):
    t, meta = golden(name)
    cfg, m = build(meta)
    T, G = meta["T"], meta["G"]
    want = t["bf16.generate_ids"]
    margins = t["bf16.generate_margins_ulps"]
    safe = G if (margins >= SAFE_MARGIN_ULPS).all() else int((margins < SAFE_MARGIN_ULPS).nonzero()[0])
    got = generate(m, t["idx1"].to(DEV), T + G, temperature=0.2, top_k=1).cpu()
    assert got.numel() == T + G
    record_parity(f"{name}.generate_ids", generated=G, tie_free_prefix=safe, ids_equal_prefix=_equal_prefix(got[T:], want[T:]))
    assert torch.equal(got[: T + safe], want[: T + safe]), f"greedy ids differ inside the tie-free prefix ({safe} steps)"
    # EOS: stop and exclude it (Q7)
    eos = meta["bf16.eos_id"]
    ge = generate(m, t["idx1"].to(DEV), T + G, temperature=0.2, top_k=1, eos_id=eos).cpu()
    if safe >= 4:
        assert torch.equal(ge, t["bf16.generate_eos_ids"])
    # batched ragged generate == one-at-a-time generate
    p0, p1 = t["idx0"][:17].to(DEV), t["idx1"].to(DEV)
    alone = [generate(m, p, p.numel() + 6, temperature=0.2, top_k=1).cpu() for p in (p0, p1)]
    both = [o.cpu() for o in generate_batch(m, [p0, p1], 6, temperature=0.2, top_k=1)]
    assert all(torch.equal(a, b) for a, b in zip(alone, both))


@pytest.mark.parametrize("name", TINY)
def test_merged_lora_matches_unmerged(golden, name):
    t, meta = golden(name)
    cfg, m = build(meta)
    idx = torch.stack([t["idx0"], t["idx1"]]).to(DEV)
    with torch.no_grad():
        a = m(idx)
        merge_lora_weights(m)
        b = m(idx)
    yard = rel_rms(t["bf16.logits_nocache"], t["fp32.logits_nocache"])
    # merging rounds W + s*B*A to bf16 once: a different (equally valid) bf16 function
    assert rel_rms(b, t["fp32.logits_merged"]) < 1.5 * yard and rel_rms(a, b) < 1.5 * yard


def test_block_intermediates(golden):
    """TinyLlama-shape block: every intermediate the reference produced (Q10 rounding points)."""
    from dualhyp_amd import ops
    from conftest import ulp_diff as ud
    t, meta = golden("block_tinyllama")
    cfg, m = build(meta)
    blk = m.transformer.h[0]
    T = meta["T"]
    x = t["x"].to(DEV)[0]

    def chk(got, key, max_ulp=1, frac=0.02):
        want = t[key].reshape(got.shape)
        u = ud(got.float().cpu(), want.float(), 1.0)
        f = (u > 0).float().mean().item()
        record_parity(f"block_tinyllama.{key}", max_ulp=u.max().item(), frac_differ=f, bit_exact_frac=1.0 - f)
        assert u.max().item() <= max_ulp and f <= frac, f"{key}: {u.max().item()} ulp, {f:.3%}"

    with torch.no_grad():
        tail = (torch.arange(T) >= T // 32 * 32).to(torch.uint8).to(DEV)   # Q11: CPU rsqrt scalar tail
        n1 = ops.rmsnorm(x, blk.norm_1.weight, cfg.norm_eps, row_tail=tail)
        chk(n1, "norm_1")
        n1r = t["norm_1"].to(DEV)[0]                       # continue from the reference's value
        chk(ops.linear(n1r, blk.attn.attn.linear.weight), "qkv_pretrained")
        qkv = blk.attn.attn(n1r)
        chk(qkv, "qkv")
        qkvr = t["qkv"].to(DEV)[0]
        cos, sin = m.build_rope_cache(x)
        G, H, hs = cfg.n_query_groups, cfg.n_head, cfg.head_size
        kc = torch.zeros((1, G, 64, hs), dtype=torch.bfloat16, device=DEV)
        vt = torch.zeros((1, G, hs, 64), dtype=torch.bfloat16, device=DEV)
        i32 = torch.int32
        q = ops.qkv_rope_cache(qkvr, cos, sin, torch.zeros(T, dtype=i32, device=DEV), torch.arange(T, dtype=i32, device=DEV),
                               kc, vt, H, G)
        assert torch.equal(q.permute(1, 0, 2).cpu(), t["q_roped"][0])
        assert torch.equal(ops.kcache_to_plain(kc)[0, :, :T].cpu(), t["k_roped"][0])
        y = ops.attn_prefill(q, kc, vt, torch.zeros(1, dtype=i32, device=DEV), torch.zeros(1, dtype=i32, device=DEV),
                             torch.tensor([T], dtype=i32, device=DEV), torch.zeros(1, dtype=i32, device=DEV), T)
        chk(y, "attn_y", max_ulp=2, frac=0.05)
        yr = t["attn_y"].to(DEV)[0]
        x1 = blk.attn.proj(yr, resid=x)
        chk(x1, "resid_1")
        x1r = t["resid_1"].to(DEV)[0]
        n2 = ops.rmsnorm(x1r, blk.norm_2.weight, cfg.norm_eps, row_tail=tail)
        chk(n2, "norm_2")
        n2r = t["norm_2"].to(DEV)[0]
        act = ops.linear(n2r, blk.mlp.fc_1.linear.weight, epilogue=ops.EPI_SWIGLU, w2=blk.mlp.fc_2.linear.weight)
        chk(act, "mlp_act", max_ulp=2, frac=0.003)
        out = blk.mlp(n2r, resid=x1r)
        chk(out, "block_out", max_ulp=2, frac=0.005)   # K = 5632 fp32 chain order vs torch's: rounding-boundary flips


@pytest.mark.parametrize("name", TINY)
def test_joint_decode_is_batch_invariant(golden, name):
    """More prompts than `prefill_batch`: chunked prefill into consecutive KV slots + ONE decode loop over
    all rows (up to 256 rows take the streaming kernels).  Every row equals the same prompt generated in
    a small batch / alone: the kernels' per-row summation order does not depend on the packing."""
    t, meta = golden(name)
    cfg, m = build(meta)
    V = cfg.padded_vocab_size
    g = torch.Generator().manual_seed(7)
    prompts = [torch.randint(3, V, (int(n),), generator=g).to(DEV) for n in torch.randint(5, 40, (70,), generator=g)]
    G = 9
    # beyond 128 rows the SwiGLU (64: lm_head) GEMMs switch to the tiled kernel (gemm_dt.hip), from 1280 rows the
    # partial-sum GEMMs too (its chain mode, consumers then see one partial) — same bits either way
    joint = [o.cpu() for o in generate_batch(m, prompts, G, temperature=0.2, top_k=1, prefill_batch=16)]
    assert len(joint) == 70
    for reps in (3, 12, 20):                                   # 210, 840 and 1400 rows
        big = [o.cpu() for o in generate_batch(m, prompts * reps, G, temperature=0.2, top_k=1, prefill_batch=64)]
        assert len(big) == 70 * reps
        assert all(torch.equal(big[i], joint[i % 70]) for i in range(70 * reps)), f"{70 * reps}-row decode differs from the 70-row decode"
    for a in range(0, 70, 16):
        part = [o.cpu() for o in generate_batch(m, prompts[a:a + 16], G, temperature=0.2, top_k=1)]
        for i, o in enumerate(part):
            assert torch.equal(o, joint[a + i]), f"row {a + i} differs between the 70-row and the 16-row decode"
    alone = generate(m, prompts[37], prompts[37].numel() + G, temperature=0.2, top_k=1).cpu()
    assert torch.equal(alone, joint[37])
    # EOS inside a joint decode: rows stop independently (device-side done flags), the EOS is excluded (Q7)
    flat = torch.cat([o[p.numel():] for o, p in zip(joint, prompts)])
    eos = int(torch.bincount(flat, minlength=V).argmax())          # a token the greedy decode produces often
    je = [o.cpu() for o in generate_batch(m, prompts, G, temperature=0.2, top_k=1, eos_id=eos, prefill_batch=16)]
    stopped = 0
    for i, (o, full, p) in enumerate(zip(je, joint, prompts)):
        gen = full[p.numel():]
        hit = (gen == eos).nonzero()
        want = full if hit.numel() == 0 else full[: p.numel() + int(hit[0])]
        stopped += hit.numel() > 0
        assert torch.equal(o, want), f"row {i}: EOS handling differs in the joint decode"
    assert stopped >= 3
    ae = generate(m, prompts[5], prompts[5].numel() + G, temperature=0.2, top_k=1, eos_id=eos).cpu()
    assert torch.equal(ae, je[5])


def test_full_tinyllama_vs_reference(golden):
    """22-layer TinyLlama-1.1B shape, weights from the hash: per-step logits of the reference's
    greedy decode (teacher-forced on its ids) and free-running ids up to the first near-tie."""
    t, meta = golden("full_tinyllama")
    cfg, m = build(meta)
    T, G = meta["T"], meta["G"]
    ids = t["generate_ids"]
    margins = t["generate_margins_ulps"]
    with torch.no_grad():
        lg = m(t["idx"].view(1, -1).to(DEV), torch.arange(T, device=DEV))
        got = [lg[0, -1]]
        u = ulp_diff(lg[0, -4:].float().cpu(), t["prefill_logits_last4"].float())
        record_parity("full_tinyllama.prefill_last4", max_ulp=u.max().item(), bit_exact_frac=(u == 0).float().mean().item(),
                      rel_rms=rel_rms(lg[0, -4:], t["prefill_logits_last4"]))
        for s in range(G - 1):
            got.append(m(ids[T + s].view(1, 1).to(DEV), torch.tensor([T + s], device=DEV))[0, 0])
    got = torch.stack(got).float().cpu()
    want = t["step_logits"].float()
    f32 = t["step_logits_fp32_v4096"].float()
    rr, yard = rel_rms(got, want), rel_rms(want[:, :4096], f32)
    e_hip, e_ref = (got[:, :4096] - f32).abs().max().item(), (want[:, :4096] - f32).abs().max().item()
    record_parity("full_tinyllama.step_logits", rel_rms_hip_vs_ref=rr, rel_rms_ref_vs_fp32=yard, max_abs_hip_vs_fp32=e_hip,
                  max_abs_ref_vs_fp32=e_ref, bit_exact_frac=(got == want).float().mean().item())
    assert rr <= yard and e_hip <= 1.5 * e_ref
    am = got.argmax(-1)
    for s in range(G):
        if margins[s] >= SAFE_MARGIN_ULPS:
            assert am[s].item() == ids[T + s].item(), f"step {s}: margin {margins[s]} ulps but argmax differs"
    m.reset_cache()
    safe = G if (margins >= SAFE_MARGIN_ULPS).all() else int((margins < SAFE_MARGIN_ULPS).nonzero()[0])
    free = generate(m, t["idx"].to(DEV), T + G, temperature=0.2, top_k=1).cpu()
    assert torch.equal(free[: T + safe], ids[: T + safe])
    # joint decode of 70 rows at full size (chunked prefill, streaming kernels with 4 row groups):
    # the golden prompt rides in row 40 and must come out exactly as alone
    g = torch.Generator().manual_seed(11)
    V = cfg.padded_vocab_size
    prompts = [torch.randint(3, V, (int(n),), generator=g).to(DEV) for n in torch.randint(8, 48, (70,), generator=g)]
    prompts[40] = t["idx"].to(DEV)
    joint = generate_batch(m, prompts, G, temperature=0.2, top_k=1, prefill_batch=32)
    assert torch.equal(joint[40].cpu(), free)
    small = generate_batch(m, prompts[32:64], G, temperature=0.2, top_k=1)
    assert all(torch.equal(a, b) for a, b in zip(small, joint[32:64]))


def _teacher_forced(m, idx, ids, T, G, fwd=None):
    """Last-position logits of the prefill and of G-1 single-token steps fed the REFERENCE's ids."""
    fwd = fwd or (lambda x, pos: m(x, pos))
    with torch.no_grad():
        lg = fwd(idx.view(1, -1).to(DEV), torch.arange(T, device=DEV))
        got = [lg[0, -1]]
        for s in range(G - 1):
            got.append(fwd(ids[T + s].view(1, 1).to(DEV), torch.tensor([T + s], device=DEV))[0, 0])
    m.reset_cache()
    return torch.stack(got).float().cpu()


def test_full_tinyllama_512_vs_reference(golden):
    """BASELINE config 2's own shape: 22-layer TinyLlama-1.1B, a 512-token prompt, 64 tokens generated by the
    REFERENCE's generate() (tests/golden/full_tinyllama_512).  The fixture's head is tied to the scaled embedding
    (dualhyp_amd.synth: embed_scale / head_tie), so the reference's top-2 margin is tens of bf16 ulps — and >= 20 sigma of
    the noise that separates two bf16 implementations — on EVERY one of the 64 steps: the ids are a property of the
    function.  Asserted: all 64 free-running greedy ids equal the reference's (alone, with the product's default
    rsqrt rounding, and as one row of a 64-row joint decode = the benchmark's schedule); teacher-forced on the
    reference's ids the arg-max agrees on every step and the logits are closer to the reference's bf16 run than that
    run is to its own fp32 run."""
    t, meta = golden("full_tinyllama_512")
    cfg, m = build(meta)
    T, G = meta["T"], meta["G"]
    ids, margins = t["generate_ids"], t["generate_margins_ulps"]
    assert G == 64 and T == 512 and float(margins.min()) >= 16, "fixture must be tie-free on every step"
    got = _teacher_forced(m, t["idx"], ids, T, G)
    agree = got.argmax(-1) == ids[T:T + G]
    want, f32 = t["step_logits_v4096"].float(), t["step_logits_fp32_v4096"].float()
    rr, yard = rel_rms(got[:, :4096], want), rel_rms(want, f32)
    e_hip, e_ref = (got[:, :4096] - f32).abs().max().item(), (want - f32).abs().max().item()
    tv, ti = t["step_top8_values"].float(), t["step_top8_indices"]
    top_u = ulp_diff(torch.gather(got, 1, ti), tv)
    # margin in units of the noise between two bf16 implementations: relRMS(ref bf16, ref fp32) x logit rms, both candidates
    sigma = yard * want.pow(2).mean().sqrt().item() * 2 ** 0.5
    margin_sigma = ((tv[:, 0] - tv[:, 1]) / sigma).min().item()
    record_parity("full_tinyllama_512.teacher_forced", steps=G, min_margin_ulps=float(margins.min()), min_margin_sigma=margin_sigma,
                  argmax_equal_steps=int(agree.sum()), rel_rms_hip_vs_ref=rr, rel_rms_ref_vs_fp32=yard, max_abs_hip_vs_fp32=e_hip,
                  max_abs_ref_vs_fp32=e_ref, bit_exact_frac_v4096=(got[:, :4096] == want).float().mean().item(),
                  top8_max_ulp=top_u.max().item(), top8_bit_exact_frac=(top_u == 0).float().mean().item())
    assert margin_sigma >= 8
    assert bool(agree.all()), f"arg-max differs on steps {(~agree).nonzero().flatten().tolist()}"
    assert rr <= yard and e_hip <= 1.5 * e_ref
    assert top_u.max().item() <= 4, f"a top-8 logit is {top_u.max().item()} ulps from the reference's"
    free = generate(m, t["idx"].to(DEV), T + G, temperature=0.2, top_k=1).cpu()
    n_eq = _equal_prefix(free[T:], ids[T:])
    record_parity("full_tinyllama_512.free_running", generated=G, ids_equal_prefix=n_eq, distinct_ids=int(ids[T:].unique().numel()))
    assert n_eq == G, f"free-running greedy ids diverge from the reference's at step {n_eq} of {G}"
    m.cpu_rsqrt_vec_width = 0            # the product default: rsqrt rounded once, what a GPU run of the reference does
    free0 = generate(m, t["idx"].to(DEV), T + G, temperature=0.2, top_k=1).cpu()
    record_parity("full_tinyllama_512.free_running_product_default", generated=G, ids_equal_prefix=_equal_prefix(free0[T:], ids[T:]))
    assert torch.equal(free0, ids)
    m.cpu_rsqrt_vec_width = 32
    # the benchmark's schedule: 64 such prompts prefilled 32 at a time, decoded jointly — row 37 is the golden prompt
    g = torch.Generator().manual_seed(3)
    V = cfg.padded_vocab_size
    prompts = [torch.cat([torch.ones(1, dtype=torch.int64), torch.randint(3, V, (T - 1,), generator=g)]).to(DEV) for _ in range(64)]
    prompts[37] = t["idx"].to(DEV)
    joint = generate_batch(m, prompts, G, temperature=0.2, top_k=1, prefill_batch=32)
    assert torch.equal(joint[37].cpu(), ids), "row 37 of the 64-row joint decode differs from the reference's ids"
    from dualhyp_amd.synth import tie_successor
    assert all(int(o[T]) == tie_successor(int(p[-1]), V) for o, p in zip(joint, prompts)), "a row's first token is not the tied successor"


def test_full_tinyllama_512_untied_vs_reference(golden):
    """The benchmark's shape with NOTHING tied (tests/golden/full_tinyllama_512_untied: plain N(0, 0.02) hash weights, 22
    layers, T = 512, 64 tokens generated by the reference's generate()).  Every logit is a context-dependent projection
    of the last hidden state, so the arg-max of a step depends on the attention over 512 + s cached keys, the MLPs and
    the KV cache of all 22 layers — this is the id evidence the tied fixture cannot give (its ids are a permutation
    chain of the last token).  The price is near-ties: the reference's own top-2 margin is below 4 bf16 ulps on some
    steps (inside its bf16 noise; exact ties are decided by its RNG, Q6), so the arg-max is asserted, teacher-forced on
    the reference's ids, on exactly the steps the reference decides by >= 4 ulps, the logits are gated on all 64 steps,
    and the free-running ids on the reference's tie-free prefix."""
    t, meta = golden("full_tinyllama_512_untied")
    cfg, m = build(meta)
    T, G = meta["T"], meta["G"]
    assert T == 512 and G == 64 and "head_tie" not in meta
    ids, margins = t["generate_ids"], t["generate_margins_ulps"]
    decided = margins >= SAFE_MARGIN_ULPS
    assert int(decided.sum()) >= 40, "fixture must decide most steps by >= 4 ulps"
    got = _teacher_forced(m, t["idx"], ids, T, G)
    agree = got.argmax(-1) == ids[T:T + G]
    want, f32 = t["step_logits_v4096"].float(), t["step_logits_fp32_v4096"].float()
    rr, yard = rel_rms(got[:, :4096], want), rel_rms(want, f32)
    e_hip, e_ref = (got[:, :4096] - f32).abs().max().item(), (want - f32).abs().max().item()
    tv, ti = t["step_top8_values"].float(), t["step_top8_indices"]
    top_u = ulp_diff(torch.gather(got, 1, ti), tv)
    # the fp32 run's arg-max: how often two exact-arithmetic-close runs agree at all on this fixture
    ref_vs_f32 = int((t["step_top8_indices"][:, 0] == t["step_top8_indices_fp32"][:, 0]).sum())
    hip_vs_f32 = int((got.argmax(-1) == t["step_top8_indices_fp32"][:, 0]).sum())
    record_parity("full_tinyllama_512_untied.teacher_forced", steps=G, steps_decided_ge4ulp=int(decided.sum()),
                  argmax_equal_on_decided_steps=int((agree & decided).sum()), argmax_equal_all_steps=int(agree.sum()),
                  ref_bf16_argmax_equals_ref_fp32=ref_vs_f32, hip_argmax_equals_ref_fp32=hip_vs_f32,
                  rel_rms_hip_vs_ref=rr, rel_rms_ref_vs_fp32=yard, max_abs_hip_vs_fp32=e_hip, max_abs_ref_vs_fp32=e_ref,
                  bit_exact_frac_v4096=(got[:, :4096] == want).float().mean().item(), top8_max_ulp=top_u.max().item(),
                  top8_bit_exact_frac=(top_u == 0).float().mean().item(), distinct_ids=int(ids[T:].unique().numel()))
    bad = (decided & ~agree).nonzero().flatten().tolist()
    assert not bad, f"arg-max differs on steps the reference decides by >= 4 ulps: {bad} (margins {[float(margins[s]) for s in bad]})"
    assert rr <= yard and e_hip <= 1.5 * e_ref
    safe = G if bool(decided.all()) else int((~decided).nonzero()[0])
    free = generate(m, t["idx"].to(DEV), T + G, temperature=0.2, top_k=1).cpu()
    n_eq = _equal_prefix(free[T:], ids[T:])
    record_parity("full_tinyllama_512_untied.free_running", generated=G, reference_tie_free_prefix=safe, ids_equal_prefix=n_eq)
    assert n_eq >= safe, f"free-running greedy ids diverge at step {n_eq}, inside the reference's tie-free prefix of {safe}"
    # batch invariance on context-dependent ids: the same prompt as row 21 of a 64-row joint decode (the benchmark's
    # schedule: prefilled 32 at a time, decoded together) gives the ids of the alone run, all 64 of them
    g = torch.Generator().manual_seed(5)
    V = cfg.padded_vocab_size
    prompts = [torch.cat([torch.ones(1, dtype=torch.int64), torch.randint(3, V, (T - 1,), generator=g)]).to(DEV) for _ in range(64)]
    prompts[21] = t["idx"].to(DEV)
    joint = generate_batch(m, prompts, G, temperature=0.2, top_k=1, prefill_batch=32)
    assert torch.equal(joint[21].cpu(), free), "row 21 of the 64-row joint decode differs from the same prompt decoded alone"


@pytest.mark.parametrize("name", ["relprompt_tiny", "relprompt_hs128"])
def test_relprompt_decoder_vs_reference(golden, name):
    """BASELINE config 4's decoder: dualhyp_amd.relprompt.GPT against tensors ger.relprompt.GPT produced
    (ger/relprompt.py:215-294): wte grown by resize_token_embeddings(3), prompts containing ids V..V+2, logits over the
    original V entries, the forward(idx, audio_query, lip_query, max_seq_length, input_pos) signature."""
    from dualhyp_amd.relprompt import GPT as RelGPT
    t, meta = golden(name)
    cfg, m = build(meta, cls=RelGPT, extra=True)
    V = cfg.padded_vocab_size
    m.resize_token_embeddings(3)
    assert m.transformer.wte.weight.shape == (V + 3, cfg.n_embd) and m.lm_head.linear.weight.size(0) == V
    m.transformer.wte.weight.data[V:] = t["wte_extra_rows"].to(DEV)
    m.refresh_engine()
    T, G = meta["T"], meta["G"]
    idx = torch.stack([t["idx0"], t["idx1"]]).to(DEV)
    assert int(idx.max()) >= V
    with torch.no_grad():
        lg = m(idx)
        assert lg.shape == (2, T, V)
        gate(lg, t["bf16.logits_nocache"], t["fp32.logits_nocache"], f"{name} no-cache logits")
        m.reset_cache()
        lp = m(t["idx0"].view(1, -1).to(DEV), None, None, None, torch.arange(T, device=DEV))
        gate(lp, t["bf16.logits_prefill"], t["fp32.logits_prefill"], f"{name} prefill logits")
        same_path = t["fp32.decode_tokens"].tolist() == t["bf16.decode_tokens"].tolist()
        for s, tok in enumerate(t["bf16.decode_tokens"].tolist()):
            ld = m(torch.tensor([[tok]], device=DEV), input_pos=torch.tensor([T + s], device=DEV))
            if same_path:    # one row of V logits: its relRMS scatters +-30 % around the many-row figure gated above
                gate(ld[0, 0], t["bf16.logits_decode"][s], t["fp32.logits_decode"][s], f"{name} decode step {s}", frac=1.5, abs_frac=2.0)
        m.reset_cache()
    margins = t["bf16.generate_margins_ulps"]
    safe = G if (margins >= SAFE_MARGIN_ULPS).all() else int((margins < SAFE_MARGIN_ULPS).nonzero()[0])
    got = generate(m, t["idx1"].to(DEV), T + G, temperature=0.2, top_k=1).cpu()
    record_parity(f"{name}.generate_ids", generated=G, tie_free_prefix=safe, ids_equal_prefix=_equal_prefix(got[T:], t["bf16.generate_ids"][T:]))
    assert torch.equal(got[: T + safe], t["bf16.generate_ids"][: T + safe])


def test_relprompt_full_size_vs_reference(golden):
    """BASELINE config 4's decoder at FULL size (tests/golden/relprompt_tinyllama: ger.relprompt.GPT with TinyLlama-1.1B's 22
    layers and the GER LoRA set, nothing tied, wte grown by three rows, a 560-token prompt carrying 56 reliability tokens,
    16 tokens by the reference's greedy loop).  Same gates as the untied benchmark-shape fixture: teacher-forced on the
    reference's ids the arg-max agrees on every step the reference decides by >= 4 bf16 ulps, the logits are closer to
    the reference's bf16 run than that run is to its own fp32 run, and the free-running ids equal the reference's inside
    its tie-free prefix."""
    from dualhyp_amd.relprompt import GPT as RelGPT
    t, meta = golden("relprompt_tinyllama")
    cfg, m = build(meta, cls=RelGPT, extra=True)
    V = cfg.padded_vocab_size
    m.resize_token_embeddings(3)
    m.transformer.wte.weight.data[V:] = t["wte_extra_rows"].to(DEV)
    m.refresh_engine()
    T, G = meta["T"], meta["G"]
    idx, ids, margins = t["idx"], t["generate_ids"], t["generate_margins_ulps"]
    assert cfg.n_layer == 22 and T == 560 and int((idx >= V).sum()) == meta["reliability_tokens"] == 56
    decided = margins >= SAFE_MARGIN_ULPS
    got = _teacher_forced(m, idx, ids, T, G, fwd=lambda x, pos: m(x, input_pos=pos))
    assert got.shape == (G, V)
    agree = got.argmax(-1) == ids[T:T + G]
    want, f32 = t["step_logits_v4096"].float(), t["step_logits_fp32_v4096"].float()
    rr, yard = rel_rms(got[:, :4096], want), rel_rms(want, f32)
    e_hip, e_ref = (got[:, :4096] - f32).abs().max().item(), (want - f32).abs().max().item()
    top_u = ulp_diff(torch.gather(got, 1, t["step_top8_indices"]), t["step_top8_values"].float())
    record_parity("relprompt_tinyllama.teacher_forced", steps=G, steps_decided_ge4ulp=int(decided.sum()),
                  argmax_equal_on_decided_steps=int((agree & decided).sum()), argmax_equal_all_steps=int(agree.sum()),
                  rel_rms_hip_vs_ref=rr, rel_rms_ref_vs_fp32=yard, max_abs_hip_vs_fp32=e_hip, max_abs_ref_vs_fp32=e_ref,
                  top8_max_ulp=top_u.max().item(), bit_exact_frac_v4096=(got[:, :4096] == want).float().mean().item())
    bad = (decided & ~agree).nonzero().flatten().tolist()
    assert not bad, f"arg-max differs on steps the reference decides by >= 4 ulps: {bad}"
    assert rr <= yard and e_hip <= 1.5 * e_ref
    safe = G if bool(decided.all()) else int((~decided).nonzero()[0])
    free = generate(m, idx.to(DEV), T + G, temperature=0.2, top_k=1).cpu()
    n_eq = _equal_prefix(free[T:], ids[T:])
    record_parity("relprompt_tinyllama.free_running", generated=G, reference_tie_free_prefix=safe, ids_equal_prefix=n_eq)
    assert n_eq >= safe


@pytest.mark.parametrize("name", ["llama3_shape", "llama3_shape_1536"])
def test_llama3_8b_shape_vs_reference(golden, name):
    """BASELINE config 5's layer shape (Llama-3-8B, ger/config.py:801-818: d 4096, 32 heads / 8 groups, hs 128,
    I 14336, V 128256, LoRA r 16) with 2 layers, bf16: prefill + decode logits and greedy ids against the reference — at
    T = 96 and at the configuration's real prompt length, T = 1536 + 16 decode steps (`llama3_shape_1536`, VERDICT r03 #2:
    24 key tiles of the prefill's online softmax, 49-50 32-key tiles per (sequence, group) in the fused decode attention)."""
    t, meta = golden(name)
    cfg, m = build(meta)
    assert (cfg.n_embd, cfg.head_size, cfg.n_query_groups, cfg.intermediate_size, cfg.padded_vocab_size) == (4096, 128, 8, 14336, 128256)
    T, G = meta["T"], meta["G"]
    ids, margins = t["generate_ids"], t["generate_margins_ulps"]
    with torch.no_grad():
        lg = m(t["idx"].view(1, -1).to(DEV), torch.arange(T, device=DEV))[0].float().cpu()
    m.reset_cache()
    gate(lg[-4:, :4096], t["prefill_logits_last4_v4096"], t["prefill_logits_last4_v4096_fp32"], f"{name} prefill logits")
    u = ulp_diff(lg[-4:, -256:], t["prefill_logits_last4_tail256"].float(), 1.0)     # ulps at max(|a|, |b|, rms)
    rr_tail = rel_rms(lg[-4:, -256:], t["prefill_logits_last4_tail256"])
    record_parity(f"{name}.prefill_vocab_tail", max_ulp=u.max().item(), bit_exact_frac=(u == 0).float().mean().item(), rel_rms=rr_tail)
    # the last 256 of the 128256 vocabulary rows (the lm_head's ragged last tile): as close to the reference as the first 4096
    assert rr_tail <= rel_rms(t["prefill_logits_last4_v4096"], t["prefill_logits_last4_v4096_fp32"])
    got = _teacher_forced(m, t["idx"], ids, T, G)
    gate(got[:, :4096], t["step_logits_v4096"], t["step_logits_fp32_v4096"], f"{name} step logits")
    assert float(margins.min()) >= 16, "fixture must be tie-free on every step"
    agree = got.argmax(-1) == ids[T:T + G]
    assert bool(agree.all())
    free = generate(m, t["idx"].to(DEV), T + G, temperature=0.2, top_k=1).cpu()
    record_parity(f"{name}.generate_ids", generated=G, min_margin_ulps=float(margins.min()), ids_equal_prefix=_equal_prefix(free[T:], ids[T:]))
    assert torch.equal(free, ids)
